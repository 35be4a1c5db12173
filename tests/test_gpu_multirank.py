"""Multi-process rehearsal of the z-slab path on ONE GPU.

RCCL refuses two ranks on one device, and the driver's 2/4/8-GPU runs are the only place the
per-rank code path (World(rank=r) + dist.hip) would otherwise execute.  Here N processes share the
box's GPU with tests/fake_rccl (a strict shared-memory stand-in for the nine RCCL calls the library
makes - TEST DOUBLE, see its header) linked under the product's own objects (libndsm_hip_fake.so): ordering, peer bookkeeping, message sizes and
deadlock-freedom of the real product code are exercised, and every rank's planes must equal the
single-domain solver's bit for bit.  What it cannot cover: RCCL/xGMI themselves."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libndsm_hip_fake.so")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import ndsm_amd
    from ndsm_amd import _lib
    L = ndsm_amd.load_library()
    assert L.ndsm_hip_init(0) == 0, _lib.last_error(L)
    return _lib


def _fake():
    r = subprocess.run(["make", "-s", "-C", os.path.dirname(FAKE)])        # no-op when up to date
    if r.returncode != 0 and not os.path.exists(FAKE):
        pytest.fail("tests/fake_rccl could not be built")
    return FAKE


def _run_world(tmp_path, world, cases, timeout=300, worker="multirank_worker.py", extra_env=None):
    out = str(tmp_path)
    json.dump(cases, open(os.path.join(out, "cases.json"), "w"))
    env = dict(os.environ, NDSM_HIP_LIB=_fake(), FAKE_RCCL_TIMEOUT="60")
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", worker), str(r), str(world), out],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs, codes = [], []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            logs.append(o)
            codes.append(p.returncode)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for i, l in enumerate(logs):
        with open(os.path.join(out, f"rank{i}.log"), "w") as f:
            f.write(l)
    assert all(c == 0 for c in codes), "\n".join(f"--- rank {i} rc {c}\n{l[-3000:]}" for i, (c, l) in enumerate(zip(codes, logs)))
    return out


def _check_against_single(hip, out, world, cases):
    from golden_inputs import rand_field, uniform_mesh
    for ci, c in enumerate(cases):
        ns = c["ns"]
        mesh = uniform_mesh(ns)
        shp = tuple(ns[::-1])
        u, rhs = rand_field(shp, 2112), rand_field(shp, 2113)
        S = hip.MGSolver(ns, mesh, c["bcs"])
        if c.get("precision"):
            assert S.set_precision(c["precision"])
        infos = [json.load(open(os.path.join(out, f"c{ci}_r{r}.json"))) for r in range(world)]
        assert infos[0]["z0"] == 0 and infos[-1]["z1"] == ns[2]
        assert all(infos[r]["z1"] == infos[r + 1]["z0"] for r in range(world - 1))
        if "levels" in c:
            assert all(i["dist_levels"] == c["levels"] for i in infos), [i["dist_levels"] for i in infos]

        def gathered(tag):
            return np.concatenate([np.load(os.path.join(out, f"c{ci}_{tag}_r{r}.npy")) for r in range(world)], axis=0)

        def fresh():
            S.upload(1, hip.BUF_U, u)
            if c.get("laplace"):
                S.zero_rhs()
            else:
                S.upload(1, hip.BUF_RHS, rhs)

        fresh()
        S.op(hip.OP_RELAX, 1, 3)
        assert np.array_equal(gathered("relax"), S.download(1, hip.BUF_U)), f"case {ci} sweeps"
        fresh()
        S.vcycle(2)
        assert np.array_equal(gathered("vcycle"), S.download(1, hip.BUF_U)), f"case {ci} vcycle"
        ierr, du, nc, hist = S.solve(hist_len=64)
        for i in infos:      # every rank sees the same all-reduced history and stops in the same cycle
            assert (i["ierr"], i["nc"]) == (ierr, nc) and i["hist"] == [float(h) for h in hist], (ci, i["nc"], nc)
        assert np.array_equal(gathered("solve"), S.download(1, hip.BUF_U)), f"case {ci} solve"
        S.close()


@pytest.mark.parametrize("world", (2, 3))
def test_ranks_on_one_gpu_bitwise(hip, tmp_path, world):
    """level 1 in slabs, levels >= 2 on rank 0; with and without the overlapped halo exchange"""
    cases = [
        {"ns": [64, 48, 96], "bcs": "NDDNDD", "env": {"NDSM_HIP_OVERLAP": "1"}},
        {"ns": [64, 48, 96], "bcs": "DDNDDN", "env": {"NDSM_HIP_OVERLAP": "0"}},
        {"ns": [67, 40, 72], "bcs": "DNDDND", "laplace": True},
    ]
    out = _run_world(tmp_path, world, cases)
    _check_against_single(hip, out, world, cases)


@pytest.mark.parametrize("world,ns,levels", ((4, [64, 64, 256], 2), (2, [128, 64, 256], 3)), ids=str)
def test_ranks_on_one_gpu_distributed_levels(hip, tmp_path, world, ns, levels):
    """several distributed levels: child worlds, restriction into the own slab of the next level"""
    cases = [{"ns": ns, "bcs": b, "levels": levels, "env": {"NDSM_HIP_DIST_LEVELS": str(levels)}} for b in ("NDDNDD", "DDNDDN")]
    out = _run_world(tmp_path, world, cases)
    _check_against_single(hip, out, world, cases)


@pytest.mark.parametrize("world", (2, 3))
def test_ranks_on_one_gpu_mixed_precision(hip, tmp_path, world):
    """mixed precision on the slabs (fp64 residual, fp32 correction V-cycle with fp32 halo exchange,
    restriction and prolongation) against the single-domain mixed mode: sweeps and V-cycles stay fp64
    operations; the solve - du history, cycle count, solution bits - must agree exactly"""
    cases = [
        {"ns": [128, 64, 96], "bcs": "NDDNDD", "precision": 2, "laplace": True},
        {"ns": [64, 80, 120], "bcs": "DDNDDN", "precision": 2},
        {"ns": [128, 64, 192], "bcs": "DNDDND", "precision": 2, "levels": 2, "env": {"NDSM_HIP_DIST_LEVELS": "2"}},
    ]
    out = _run_world(tmp_path, world, cases)
    _check_against_single(hip, out, world, cases)


@pytest.mark.parametrize("world", (2, 3))
def test_distributed_vector_potential_bitwise(hip, tmp_path, world):
    """ndsm_hip_world_vector_solve (ndsmh_wvecpot: faces gathered on rank 0, 3-D solves on z-slab worlds,
    flux balance + curl on the slabs) against ndsm_vector_solve on the whole field: every rank's planes
    of A and B, bit for bit - analytic and unbalanced boundary data, non-zero initial guess, the
    curl-first option order is covered by the same kernels"""
    import ndsm_amd
    from golden_inputs import analytic_case
    cases = [
        {"ns": [40, 36, 48]},
        {"ns": [33, 30, 45], "noise": 5, "guess": 6, "kw": {"ms": 3, "mean": True}},
        {"ns": [64, 48, 96], "noise": 7, "kw": {"ncycles_max": 3}},
        {"ns": [64, 64, 96], "kw": {"mixed_precision": 2}},
    ]
    out = _run_world(tmp_path, world, cases, worker="multirank_vecpot_worker.py")
    for ci, c in enumerate(cases):
        x, y, z, A1, b = analytic_case(c["ns"])
        if c.get("noise"):
            b = b + 0.3 * np.random.default_rng(c["noise"]).uniform(-1, 1, b.shape)
        kw = dict(c.get("kw", {}))
        if c.get("guess"):
            a0 = np.random.default_rng(c["guess"]).uniform(-1, 1, b.shape)
            ierr, A, B = ndsm_amd.vector_potential_slab(x, y, z, b, 0, 1, a_init=a0, **kw)     # nranks = 1: the plain call
        else:
            ierr, A, B = ndsm_amd.vector_potential(x, y, z, b.copy(), **kw)
        infos = [json.load(open(os.path.join(out, f"v{ci}_r{r}.json"))) for r in range(world)]
        assert all(i["ierr"] == ierr for i in infos), (ierr, infos)
        gA = np.concatenate([np.load(os.path.join(out, f"v{ci}_A_r{r}.npy")) for r in range(world)], axis=1)
        gB = np.concatenate([np.load(os.path.join(out, f"v{ci}_B_r{r}.npy")) for r in range(world)], axis=1)
        assert np.array_equal(gA, A), (ci, np.abs(gA - A).max())
        assert np.array_equal(gB, B), (ci, np.abs(gB - B).max())


def test_baseline_config4_workload_four_ranks(hip, tmp_path):
    """BASELINE config[4] AS A WORKLOAD - the distributed vector-potential pipeline (ndsm_hip_world_vector_solve: faces
    on rank 0, three 3-D solves on z-slab worlds, flux balance + curl on the slabs) in its mixed fp32-smoother /
    fp64-residual mode - at 640 x 640 x 320 (131 M points, 1 GiB per fp64 array) on four processes over the test
    double, against ndsm_vector_solve in the same mode on the whole field: every rank's planes of A and B bit for
    bit, all three components iterating.  (Four ranks: the pool's process guard allows six GPU processes on a box.
    The grid of config[4] itself, 2048 x 2048 x 1024, runs in this mode as ONE component on one GPU against eight
    loop-back slabs in test_gpu_parity.py::test_baseline_config4_full_grid_mixed_component; what stays 8-GPU-only is
    the whole pipeline at that size - 3 x 32 GiB of A and of B next to a hierarchy - and real RCCL / xGMI.)"""
    import ndsm_amd
    from golden_inputs import analytic_case
    cases = [{"ns": [640, 640, 320], "noise": 11, "kw": {"mixed_precision": 1}}]
    out = _run_world(tmp_path, 4, cases, timeout=1200, worker="multirank_vecpot_worker.py",
                     extra_env={"FAKE_RCCL_TIMEOUT": "300", "FAKE_RCCL_SLOT_MB": "32"})
    c = cases[0]
    x, y, z, _A1, b = analytic_case(c["ns"])
    b = b + 0.3 * np.random.default_rng(c["noise"]).uniform(-1, 1, b.shape)
    ierr, A, B = ndsm_amd.vector_potential(x, y, z, b, **c["kw"])
    del b
    infos = [json.load(open(os.path.join(out, f"v0_r{r}.json"))) for r in range(4)]
    assert all(i["ierr"] == ierr for i in infos), (ierr, infos)
    for name, want in (("A", A), ("B", B)):
        for r in range(4):
            got = np.load(os.path.join(out, f"v0_{name}_r{r}.npy"), mmap_mode="r")
            z0, z1 = infos[r]["z0"], infos[r]["z1"]
            assert np.array_equal(got, want[:, z0:z1]), (name, r)
            os.remove(os.path.join(out, f"v0_{name}_r{r}.npy"))


def test_bench_two_ranks_rehearsal(hip):
    """the driver's N=2 command line (torch.distributed.run, gloo rendezvous, slab world, self-check,
    ONE JSON line from rank 0), both ranks on this box's GPU over the test double"""
    import socket
    with socket.socket() as sk:                      # a port nobody is listening on
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, NDSM_HIP_LIB=_fake(), FAKE_RCCL_TIMEOUT="120", FAKE_RCCL_SLOT_MB="64")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0
    assert "z-slabs" in out["config"]["parallelism"], out["config"]
    assert out["slab_check"].startswith("bit-identical") and "MISMATCH" not in out["slab_check"], out["slab_check"]
    assert out["slab_mode"] is True and out["rccl_ranks"] == 2
    assert "fake_rccl" in out["rocm_stack"]["rccl"]
    one = out["same_workload_on_one_gpu"]               # both ranks share this GPU: the split cannot be faster
    assert one["ms_per_step"] > 0 and 0.2 < one["speedup"] < 1.3, one


def test_bench_lone_process_starts_its_own_ranks(hip):
    """`python3 bench.py --gpus N` WITHOUT a launcher (the form of the driver's N = 1 command): the process starts
    its own N ranks before touching the GPU, relays rank 0's line and exits with the launcher's code.  Four ranks
    at a reduced shape - two distributed levels, 48-plane slabs - on this box's one GPU over the test double
    (the pool's process guard allows six GPU processes on a box, this test process being one of them: an
    eight-process rehearsal cannot run here; the N = 8 plan itself - 64-plane slabs, two distributed levels -
    runs as eight loop-back slabs in test_gpu_parity.py::test_full_size_config4_loopback)."""
    env = dict(os.environ, NDSM_HIP_LIB=_fake(), FAKE_RCCL_TIMEOUT="120", FAKE_RCCL_SLOT_MB="64")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
           "--slab-shape", "256,256,192", "--no-e2e"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 4 and out["slab_mode"] is True and out["rccl_ranks"] == 4
    assert "MISMATCH" not in out["slab_check"], out["slab_check"]
    assert "chosen by a 3-cycle trial" in out["slab_check"], out["slab_check"]     # both schedules passed: the faster one is timed
    assert "REHEARSAL SHAPE" in out["config"]["workload"]
    assert out["roofline"]["frac"] is None and out["roofline"]["frac_algorithmic"] > 0     # no counter figure in slab mode


@pytest.mark.gpu
def test_bench_restarts_without_overlap_when_the_overlapped_exchange_wedges(hip):
    """a wedged overlapped exchange (simulated: the first worker of every rank never returns from its overlapped
    self-check) must still end in a result line: each rank's watchdog ends its worker, each rank's supervisor
    starts a second one with NDSM_HIP_OVERLAP=0 on a rendezvous of its own, and the line says which schedule was
    timed.  Two ranks on the test double."""
    env = dict(os.environ, NDSM_HIP_LIB=_fake(), FAKE_RCCL_TIMEOUT="120", FAKE_RCCL_SLOT_MB="64",
               NDSM_BENCH_FAKE_HANG="1", NDSM_BENCH_OVERLAP_TIMEOUT="8")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NDSM_HIP_OVERLAP"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--slab-shape", "128,128,192", "--no-e2e"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["slab_mode"] is True and out["rccl_ranks"] == 2
    assert "TIMED WITH NDSM_HIP_OVERLAP=0" in out["slab_check"] and "did not come back" in out["slab_check"]
    assert "starting it again with NDSM_HIP_OVERLAP=0" in r.stderr
