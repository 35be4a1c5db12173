"""One rank of the multi-process rehearsal (tests/test_gpu_multirank.py starts N of these on the one
GPU of the box, with NDSM_HIP_LIB pointing at the build linked against tests/fake_rccl in place of RCCL).  Runs the product's per-rank code
path - World(rank=r): slab kernels, halo exchange, restriction/prolongation across ranks, metric
all-reduce - and leaves its owned planes in <outdir> for the parent to compare with the
single-domain solver.  usage: multirank_worker.py rank world outdir"""
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
from golden_inputs import rand_field, uniform_mesh  # noqa: E402


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
        for k, v in c.get("env", {}).items():
            os.environ[k] = v
        mesh = uniform_mesh(ns)
        shp = tuple(ns[::-1])
        u, rhs = rand_field(shp, 2112), rand_field(shp, 2113)
        W = _lib.World(ns, mesh, c["bcs"], world, rank, lib=L)
        assert W.nlocal == 1
        if c.get("precision"):
            assert W.set_precision(c["precision"])
        sl = W.slabs[0]
        res = {"z0": sl["z0"], "z1": sl["z1"], "dist_levels": W.dist_levels}

        def owned():
            return W.download(_lib.BUF_U)[sl["z0"]:sl["z1"]].copy()

        W.upload(_lib.BUF_U, u)
        if c.get("laplace"):
            W.zero_rhs()
        else:
            W.upload(_lib.BUF_RHS, rhs)
        W.relax(3)
        np.save(os.path.join(out, f"c{ci}_relax_r{rank}.npy"), owned())
        W.upload(_lib.BUF_U, u)
        W.vcycle(2)
        np.save(os.path.join(out, f"c{ci}_vcycle_r{rank}.npy"), owned())
        ierr, du, nc, hist = W.solve(hist_len=64)
        np.save(os.path.join(out, f"c{ci}_solve_r{rank}.npy"), owned())
        res.update(ierr=int(ierr), du=float(du), nc=int(nc), hist=[float(h) for h in hist])
        json.dump(res, open(os.path.join(out, f"c{ci}_r{rank}.json"), "w"))
        W.close()
        for k in c.get("env", {}):
            os.environ.pop(k, None)
    _lib.dist_finalize(L)
    print(f"rank {rank}: done", flush=True)


if __name__ == "__main__":
    main()
