#!/usr/bin/env python3
"""bench.py - the hot path's headline metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 512]

Workload (BASELINE.json: "fine-grid LUP/s + V-cycles/sec at 512^3 fp64"):
config[2], the 512^3 vector-potential problem on ONE GPU - one STEP is one
multigrid V-cycle (ms = 5, reference level rule: 8 grids) of its Ax Laplace
component, boundary data from the analytic field of the reference's
integration test, arrays resident in HBM before the timed region starts.

value = fine-grid lattice-point updates per second over the WHOLE V-cycle:
        2*ms*nx*ny*nz smoother updates on level 1 per cycle / time per cycle
        (coarse levels, residual, transfers and the metric are all inside the
        timed region and count as overhead, not as updates).
Extra keys: vcycles_per_s, smoother (kernel-only, HIP events), roofline of the
dominant kernel (level-1 RB-GS sweep, 24 B/LUP algorithmic), cpu_baseline (the
reference's own smoother timed on this box's host cores).

One process per GPU.  N > 1: see DESIGN.md section "multi-GPU".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BYTES_PER_LUP = 24.0       # u read + rhs read + u write per full red+black sweep


def boundary_problem(n):
    """Ax of the analytic field (integration_test1.py:57-99) on the four
    Dirichlet faces of the Ax problem ("NDDNDD"), zero elsewhere, rhs = 0."""
    x = np.linspace(0.0, 1.0, n)
    dx = x[1] - x[0]
    y = np.arange(n) * dx
    z = np.arange(n) * dx
    wn = np.pi
    l = np.sqrt(2 * wn ** 2)
    u = np.zeros((n, n, n))                       # (nz, ny, nx)
    ax = lambda X, Y, Z: -np.cos(wn * X) * np.sin(wn * Y) * np.exp(-l * Z)  # noqa: E731
    Zg, Xg = np.meshgrid(z, x, indexing="ij")
    u[:, 0, :] = ax(Xg, y[0], Zg)
    u[:, -1, :] = ax(Xg, y[-1], Zg)
    Yg, Xg = np.meshgrid(y, x, indexing="ij")
    u[0, :, :] = ax(Xg, Yg, z[0])
    u[-1, :, :] = ax(Xg, Yg, z[-1])
    return [x, y, z], u


def cpu_baseline(seconds_budget=20.0):
    """Reference smoother (red_black_gauss_3D, ndsm_optimized.f90:40) on this
    box's host cores; falls back to the C port if oracle/_ref is absent."""
    from oracle import Oracle, have_ref, usable_cpus
    kind = "reference" if have_ref() else "port"
    orc = Oracle("ref" if have_ref() else "port")
    n = 256
    mesh = [np.linspace(0, 1, n)] * 3
    rng = np.random.default_rng(2112)
    u = rng.uniform(-1, 1, (n, n, n))
    rhs = np.random.default_rng(2113).uniform(-1, 1, (n, n, n))
    u = orc.relax3d(u, rhs, mesh, "NDDNDD")       # warm-up (thread team, page faults)
    sweeps, t0 = 0, time.perf_counter()
    while True:
        u = orc.relax3d(u, rhs, mesh, "NDDNDD")
        sweeps += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or sweeps >= 200:
            break
    # each call copies u once in the ctypes wrapper; time that copy and subtract
    t1 = time.perf_counter()
    for _ in range(3):
        _ = u.copy()
    copy_t = (time.perf_counter() - t1) / 3
    per_sweep = el / sweeps - copy_t
    return {"value": n ** 3 / per_sweep, "unit": "LUP/s", "cores": orc.threads, "kind": kind,
            "sample": f"{sweeps} sweeps of red_black_gauss_3D at {n}^3, BCs NDDNDD, "
                      f"OMP_NUM_THREADS={orc.threads} ({usable_cpus(10**6)} usable host CPUs)",
            "ms_per_sweep": per_sweep * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=512, help="points per dimension of the fine grid")
    ap.add_argument("--ms", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import ndsm_amd
    from ndsm_amd import _lib
    L = ndsm_amd.load_library()
    rc = L.ndsm_hip_init(local_rank)
    if rc != 0:
        raise SystemExit("libndsm_hip: " + _lib.last_error(L))

    def barrier_sync():
        _lib._check(L.ndsm_hip_sync(), "sync", L)
        if dist is not None:
            dist.barrier()

    n, ms = args.n, args.ms
    mesh, u0 = boundary_problem(n)
    S = _lib.MGSolver([n, n, n], mesh, "NDDNDD", ms=ms)
    S.upload(1, _lib.BUF_U, u0)
    S.upload(1, _lib.BUF_RHS, np.zeros_like(u0))
    del u0
    npts = float(n) ** 3

    # ---- the timed region: K V-cycles ------------------------------------
    S.vcycle(args.warmup)
    barrier_sync()
    t0 = time.perf_counter()
    S.vcycle(args.steps)
    barrier_sync()
    el = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([el], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t[0])
    ms_per_step = el / args.steps * 1e3

    # ---- dominant kernel: level-1 smoother sweeps under HIP events ---------
    nsw = 20
    S.op(_lib.OP_RELAX, 1, 2)
    S.sync()
    sm_ms = S.timed(lambda: S.op(_lib.OP_RELAX, 1, nsw)) / nsw
    achieved = BYTES_PER_LUP * npts / (sm_ms * 1e-3) / 1e9
    # residual + V-cycle breakdown helpers
    rs_ms = S.timed(lambda: [S.op(_lib.OP_RESIDUAL, 1) for _ in range(5)]) / 5
    sweeps, unconv = S.info()
    S.close()

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("smoother_sweep_bytes_per_launch")
        except Exception:
            traffic = None

    if rank != 0:
        return
    out = {
        "metric": "fine-grid LUP/s (RB-GS smoother updates per second of V-cycle time, 512^3 fp64)",
        "value": 2 * ms * npts * world / (ms_per_step * 1e-3),
        "unit": "LUP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{n}^3 vector-potential Ax component, one V-cycle per step (ms={ms}, "
                               f"{S.ngrids} grids), config[2] of BASELINE.json",
                   "global_points": int(npts) * world,
                   "parallelism": "single GPU" if world == 1 else f"{world} independent replicas"},
        "vcycles_per_s": world / (ms_per_step * 1e-3),
        "smoother": {"ms_per_sweep": sm_ms, "LUPs_per_s": npts / (sm_ms * 1e-3), "residual_ms": rs_ms},
        "coarse_exact_sweeps_per_cycle": sweeps / max(1, args.steps + args.warmup),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "level-1 RB-GS sweep (red+black)", "bytes_per_lup": BYTES_PER_LUP},
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
