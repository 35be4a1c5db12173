"""CPU tests of the multi-GPU (z-slab) path: the plan the library derives and a
world_size-2 run of the exchange scheme over gloo."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from golden_inputs import rand_field, uniform_mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _taps(qf, qc):
    """restriction taps [lo, hi) and prolongation brackets per the reference rule
    (ndsm_interp.f90:218-252, :373-435), 0-based - recomputed here in numpy"""
    hc = qc[1] - qc[0]
    hf = qf[1] - qf[0]
    nf = len(qf)

    def bracket(q, c):
        if c <= q[0]:
            return 0, -1
        if c >= q[-1]:
            return len(q) - 2, +1
        return min(int(np.floor((c - q[0]) / (q[1] - q[0]))), len(q) - 2), 0
    taps = []
    for c0 in qc:
        lo, side = bracket(qf, c0 - hc)
        first = lo if side < 0 else lo + 1
        lo, side = bracket(qf, c0 + hc)
        last = lo + 1 if side > 0 else lo
        taps.append((first, last + 1))
    plo = [bracket(qc, c)[0] for c in qf]
    return taps, plo, hf, nf


@pytest.mark.parametrize("ns,nranks", (([64, 64, 64], 2), ([128, 96, 160], 3), ([512, 512, 512], 8),
                                        ([1024, 1024, 512], 8), ([1024, 1024, 512], 4), ([200, 100, 70], 2)))
def test_slab_plan_properties(port, ns, nranks):
    import ndsm_amd
    mesh = uniform_mesh(ns)
    plan = ndsm_amd.slab_plan(ns, mesh, nranks)
    shapes, meshes = port.hierarchy(ns, mesh)
    nz, nzc = ns[2], int(shapes[1][2])
    taps, plo, _, _ = _taps(meshes[0][2], meshes[1][2])
    # owned planes tile [0, nz); coarse ownership tiles [0, nzc)
    assert plan[0]["z0"] == 0 and plan[-1]["z1"] == nz
    assert all(plan[r]["z1"] == plan[r + 1]["z0"] for r in range(nranks - 1))
    owned = sorted((p["ck0"], p["ck1"]) for p in plan if p["ck1"] > p["ck0"])
    assert owned[0][0] == 0 and owned[-1][1] == nzc
    assert all(owned[i][1] == owned[i + 1][0] for i in range(len(owned) - 1))
    g = plan[0]["g"]
    assert g >= 2 and all(p["g"] == g for p in plan)
    for p in plan:
        assert p["nloc"] == p["z1"] - p["z0"] + 2 * g and p["k0"] == p["z0"] - g
        assert p["z1"] - p["z0"] >= g                       # a slab can fill its neighbour's ghosts
        # every restriction tap of an owned coarse plane is inside the local window
        for K in range(p["ck0"], p["ck1"]):
            lo, hi = taps[K]
            assert lo >= p["z0"] - g and hi <= p["z1"] + g, (p, K, lo, hi)
        # the prolongation of every owned fine plane finds both coarse planes in [pk0, pk1)
        for k in range(p["z0"], p["z1"]):
            assert p["pk0"] <= plo[k] and plo[k] + 1 < p["pk1"]
        assert p["cb0"] <= min(p["pk0"], p["ck0"] if p["ck1"] > p["ck0"] else p["pk0"])
        assert p["cb1"] >= max(p["pk1"], p["ck1"])


def test_slab_plan_rejects_thin_slabs():
    import ndsm_amd
    ns = [64, 64, 32]
    with pytest.raises(ndsm_amd.NdsmHipError):
        ndsm_amd.slab_plan(ns, uniform_mesh(ns), 8)         # 4 planes per rank < ghost depth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("ns,bcs", (([24, 20, 32], "NDDNDD"), ([24, 20, 32], "DDNDDN"), ([22, 26, 40], "DNDDND")))
def test_gloo_two_rank_sweeps(port, ns, bcs):
    """world_size 2 over gloo: the slab scheme (one 2-plane exchange per sweep, red update of the
    first ghost plane recomputed, global colouring) reproduces the undecomposed sweeps bit for bit."""
    nsweeps = 3
    mesh = uniform_mesh(ns)
    shp = tuple(ns[::-1])
    want = rand_field(shp, 2112)
    rhs = rand_field(shp, 2113)
    for _ in range(nsweeps):
        want = port.relax3d(want, rhs, mesh, bcs)
    with tempfile.TemporaryDirectory() as td:
        mport = _free_port()
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(mport), OMP_NUM_THREADS="1")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "slab_gloo_worker.py"),
                                           "x".join(str(v) for v in ns), bcs, str(nsweeps), td], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = [p.communicate(timeout=180)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)
        got = np.full(shp, np.nan)
        for r in range(2):
            z0, z1, g = (int(v) for v in np.load(os.path.join(td, f"plan_{r}.npy")))
            got[z0:z1] = np.load(os.path.join(td, f"slab_{r}.npy"))
    assert np.array_equal(got, want)
