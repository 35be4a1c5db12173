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
        (coarse levels, residual, transfers, the convergence metric of update_u
        and the host's read-back of it are all inside the timed region and count
        as overhead, not as updates).
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


def cpu_baseline(seconds_budget=12.0):
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
    u = np.ascontiguousarray(u, dtype=np.float64)
    u = orc.relax3d(u, rhs, mesh, "NDDNDD", inplace=True)       # warm-up (thread team, page faults)
    sweeps, t0 = 0, time.perf_counter()
    while True:
        orc.relax3d(u, rhs, mesh, "NDDNDD", inplace=True)      # in place, as the reference's solver calls it
        sweeps += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or sweeps >= 2000:
            break
    per_sweep = el / sweeps
    out = {"value": n ** 3 / per_sweep, "unit": "LUP/s", "cores": orc.threads, "kind": kind,
           "sample": f"{sweeps} sweeps of red_black_gauss_3D at {n}^3, BCs NDDNDD, "
                     f"OMP_NUM_THREADS={orc.threads} ({usable_cpus(10**6)} usable host CPUs)",
           "ms_per_sweep": per_sweep * 1e3}
    # SURVEY 8d also asks for s/V-cycle: ONE pass of the reference's solve loop (V-cycle + update_u) at
    # 128^3 - its generic restriction / interpolation dominate it, so this is context
    # for whole-solve ratios, not a smoother comparison
    try:
        m = 128
        mesh = [np.linspace(0, 1, m)] * 3
        u0 = np.random.default_rng(7).uniform(-1, 1, (m, m, m))
        t0 = time.perf_counter()
        orc.solve_bvp(u0, np.zeros_like(u0), mesh, "NDDNDD", ms=5, nmax=1)
        out["vcycle_s"] = time.perf_counter() - t0
        out["vcycle_sample"] = f"one V-cycle + update_u of solve_poisson_bvp at {m}^3 (ms=5), same threads"
    except Exception as exc:  # noqa: BLE001
        out["vcycle_sample"] = f"not timed: {type(exc).__name__}: {exc}"
    return out


def slab_window_problem(n3, sl):
    """window [k0, k0+nloc) ∩ [0, nz) of the Ax boundary problem on an (nx, ny, nz) box (equal spacing)"""
    nx, ny, nz = n3
    x = np.linspace(0.0, 1.0, nx)
    dx = x[1] - x[0]
    y = np.arange(ny) * dx
    z = np.arange(nz) * dx
    a, b = max(sl["k0"], 0), min(sl["k0"] + sl["nloc"], nz)
    wn = np.pi
    l = np.sqrt(2 * wn ** 2)
    ax = lambda X, Y, Z: -np.cos(wn * X) * np.sin(wn * Y) * np.exp(-l * Z)  # noqa: E731
    u = np.zeros((b - a, ny, nx))
    Zg, Xg = np.meshgrid(z[a:b], x, indexing="ij")
    u[:, 0, :] = ax(Xg, y[0], Zg)
    u[:, -1, :] = ax(Xg, y[-1], Zg)
    Yg, Xg = np.meshgrid(y, x, indexing="ij")
    if a == 0:
        u[0] = ax(Xg, Yg, z[0])
    if b == nz:
        u[-1] = ax(Xg, Yg, z[-1])
    return [x, y, z], u, a


def slab_self_check(_lib, L, dist, rank, world):
    """The distributed path against the single-GPU solver on a problem small enough to gather:
    level 1 of a 128 x 128 x 32N box cut into N z-slabs (RCCL halo exchange, the same code path as
    the timed workload, two distributed levels forced), 2 V-cycles, owned planes gathered to rank 0
    over gloo and compared BIT FOR BIT with the same V-cycles of one MGSolver.  Returns a string."""
    import torch
    ns = [128, 128, 32 * world]
    dx = 1.0 / (ns[0] - 1)
    mesh = [np.arange(n) * dx for n in ns]
    rng = np.random.default_rng(2112)
    u = rng.uniform(-1, 1, tuple(ns[::-1]))
    rhs = rng.uniform(-1, 1, tuple(ns[::-1]))
    old = os.environ.get("NDSM_HIP_DIST_LEVELS")
    os.environ["NDSM_HIP_DIST_LEVELS"] = "2"
    try:
        W = _lib.World(ns, mesh, "NDDNDD", world, rank, lib=L)
    finally:
        if old is None:
            os.environ.pop("NDSM_HIP_DIST_LEVELS", None)
        else:
            os.environ["NDSM_HIP_DIST_LEVELS"] = old
    sl = W.slabs[0]
    a, b = max(sl["k0"], 0), min(sl["k0"] + sl["nloc"], ns[2])
    W.upload_window(1, _lib.BUF_U, u[a:b], a)
    W.upload_window(1, _lib.BUF_RHS, rhs[a:b], a)
    W.vcycle(2)
    mine = np.empty((sl["z1"] - sl["z0"], ns[1], ns[0]))
    _lib._check(L.ndsm_hip_world_download(W.h, 1, _lib.BUF_U, mine.ctypes.data_as(_lib._dp)), "download", L)
    levels = W.dist_levels
    W.close()
    parts = [None] * world
    dist.all_gather_object(parts, (sl["z0"], sl["z1"], mine))
    if rank != 0:
        return ""
    got = np.empty_like(u)
    for z0, z1, p in parts:
        got[z0:z1] = p
    S = _lib.MGSolver(ns, mesh, "NDDNDD", lib=L)
    S.upload(1, _lib.BUF_U, u)
    S.upload(1, _lib.BUF_RHS, rhs)
    S.vcycle(2)
    want = S.download(1, _lib.BUF_U)
    S.close()
    nd = int((got != want).sum())
    tag = f"{ns[0]}x{ns[1]}x{ns[2]} in {world} z-slabs over RCCL, {levels} distributed levels, 2 V-cycles"
    if nd == 0:
        return "bit-identical to the single-GPU solver (" + tag + ")"
    return f"MISMATCH: {nd} points differ, max {np.abs(got - want).max():.3e} (" + tag + ")"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=512, help="points per dimension of the fine grid (1 GPU)")
    ap.add_argument("--ms", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # libndsm_hip FIRST: it loads /opt/rocm's libamdhip64.so.7 / librccl.so.1.  PyTorch bundles a
    # second ROCm stack under the same SONAMEs; whichever is loaded first serves the whole process,
    # so torch is imported afterwards and used for its CPU (gloo) rendezvous only - the device is
    # synchronised through the library's own stream (ndsm_hip_sync), not torch.cuda.synchronize().
    import ndsm_amd
    from ndsm_amd import _lib
    L = ndsm_amd.load_library()
    rc = L.ndsm_hip_init(local_rank % max(1, L.ndsm_hip_device_count()))
    if rc != 0:
        raise SystemExit("libndsm_hip: " + _lib.last_error(L))

    dist = None
    rccl_ok, rccl_err, slab_mode = True, "", False
    slab_check = None
    if world > 1:
        import faulthandler
        faulthandler.dump_traceback_later(600, exit=True)   # a wedged collective must not hang the node
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(_lib.dist_unique_id(L)), dtype=torch.uint8).clone()
        dist.broadcast(uid, 0)
        try:
            _lib.dist_init(rank, world, uid.numpy().tobytes(), L)     # RCCL communicator over xGMI
        except Exception as exc:  # noqa: BLE001
            rccl_ok, rccl_err = False, f"{type(exc).__name__}: {exc}"

    def barrier_sync():
        _lib._check(L.ndsm_hip_sync(), "sync", L)
        if dist is not None:
            dist.barrier()

    ms = args.ms
    if world == 1:
        n = args.n
        n3 = [n, n, n]
        mesh, u0 = boundary_problem(n)
        S = _lib.MGSolver(n3, mesh, "NDDNDD", ms=ms)
        S.upload(1, _lib.BUF_U, u0)
        S.zero_rhs()       # the vector potential's 3-D problems are Laplace problems (rhs = 0, :640-641)
        del u0
        # one step = one pass of the reference's solve loop (solve_poisson_bvp, ndsm_poisson.f90:104-150):
        # V-cycle + update_u's max|u_new - u_old| + the host's strict du < vc_tol test; vc_tol = 0 is never
        # met, so exactly K cycles run
        run_cycles = lambda k: S.solve(vc_tol=0.0, nmax=k)  # noqa: E731
        ngrids = S.ngrids
        workload = (f"{n}^3 vector-potential Ax component, one V-cycle + convergence metric per step (ms={ms}, "
                    f"{ngrids} grids), config[2] of BASELINE.json")
        parallelism = "single GPU"
        scaling = "weak"
    else:
        # BASELINE config[3]: 1024 x 1024 x 512 Poisson, level 1 z-slab decomposed over the ranks,
        # RCCL halo exchange between sweeps, levels >= 2 on rank 0
        n3 = [1024, 1024, 512]
        x = np.linspace(0.0, 1.0, n3[0])
        dx = x[1] - x[0]
        mesh = [x, np.arange(n3[1]) * dx, np.arange(n3[2]) * dx]
        S, err = None, ""
        try:
            if not rccl_ok:
                raise RuntimeError(rccl_err)
            S = _lib.World(n3, mesh, "NDDNDD", world, rank, ms=ms, lib=L)
            _m, win, a = slab_window_problem(n3, S.slabs[0])
            S.upload_window(1, _lib.BUF_U, win, a)
            S.zero_rhs()                                          # Laplace problem, as on one GPU
            del win
            S.vcycle(1)
            S.sync()
        except Exception as exc:  # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"
            S = None
        import torch
        flag = torch.tensor([1.0 if S is not None else 0.0], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag[0]) > 0.5:
            run_cycles = lambda k: S.solve(vc_tol=0.0, nmax=k)  # noqa: E731
            ngrids = 8
            workload = (f"1024x1024x512 Poisson (Ax boundary data), one V-cycle + convergence metric per step "
                        f"(ms={ms}), config[3] of BASELINE.json")
            parallelism = (f"level 1 in {world} z-slabs (RCCL send/recv halo: 4 planes per neighbour per two-sweep "
                           "pass), levels>=2 on rank 0")
            scaling = "strong"
            slab_mode = True
            try:
                slab_check = slab_self_check(_lib, L, dist, rank, world)
            except Exception as exc:  # noqa: BLE001
                slab_check = f"self-check could not run: {type(exc).__name__}: {exc}"
        else:
            # the distributed path could not be brought up on this node: say so and measure
            # independent replicas of the 1-GPU workload instead of reporting nothing
            if S is not None:
                S.close()
            if rank == 0:
                print("bench: z-slab RCCL path unavailable (" + (err or "failed on another rank") +
                      "); falling back to independent replicas", file=sys.stderr, flush=True)
            n = args.n
            n3 = [n, n, n]
            mesh, u0 = boundary_problem(n)
            S = _lib.MGSolver(n3, mesh, "NDDNDD", ms=ms, lib=L)
            S.upload(1, _lib.BUF_U, u0)
            S.zero_rhs()
            del u0
            run_cycles = lambda k: S.solve(vc_tol=0.0, nmax=k)  # noqa: E731
            ngrids = S.ngrids
            workload = f"{n}^3 vector-potential Ax component per GPU, one V-cycle + convergence metric per step (ms={ms})"
            parallelism = f"{world} independent replicas (no exchange) - z-slab RCCL path failed to start"
            scaling = "weak"
            slab_mode = False
    npts = float(n3[0]) * n3[1] * n3[2]
    if world > 1 and not slab_mode:
        npts *= world           # every replica updates its own grid

    # ---- the timed region: K V-cycles ------------------------------------
    run_cycles(args.warmup)
    barrier_sync()
    t0 = time.perf_counter()
    run_cycles(args.steps)
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
    vc_only_ms = None
    if world == 1:
        S.vcycle(2)
        S.sync()
        vc_only_ms = S.timed(lambda: S.vcycle(args.steps)) / args.steps   # the cycle without update_u
        # Laplace variant first (what the V-cycles above ran: rhs never read, 16 B/LUP algorithmic) ...
        S.op(_lib.OP_RELAX, 1, 2)
        S.sync()
        lap_ms = S.timed(lambda: S.op(_lib.OP_RELAX, 1, nsw)) / nsw
        # ... then the general kernel with a right-hand side in HBM: the roofline line (24 B/LUP)
        S.upload(1, _lib.BUF_RHS, np.random.default_rng(2113).uniform(-1, 1, (n, n, n)))
        S.op(_lib.OP_RELAX, 1, 2)
        S.sync()
        sm_ms = S.timed(lambda: S.op(_lib.OP_RELAX, 1, nsw)) / nsw
        rs_ms = S.timed(lambda: [S.op(_lib.OP_RESIDUAL, 1) for _ in range(5)]) / 5
        sweeps, unconv = S.info()
        sweeps_per_cycle = sweeps / max(1, args.steps + args.warmup)
    elif not slab_mode:
        S.op(_lib.OP_RELAX, 1, 2)
        S.sync()
        sm_ms = S.timed(lambda: S.op(_lib.OP_RELAX, 1, nsw)) / nsw / world   # whole-job time per global sweep
        rs_ms = None
        lap_ms = None
        sweeps_per_cycle = None
    else:
        S.relax(2)
        barrier_sync()
        sm_ms = S.timed(lambda: S.relax(nsw)) / nsw               # includes the halo exchanges
        if dist is not None:
            import torch
            t = torch.tensor([sm_ms], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sm_ms = float(t[0])
        rs_ms = None
        lap_ms = None
        sweeps_per_cycle = None
    achieved = BYTES_PER_LUP * npts / (sm_ms * 1e-3) / 1e9         # whole job
    S.close()

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath) and world == 1:
        try:
            traffic = json.load(open(tpath)).get("smoother_sweep_bytes_per_launch")
        except Exception:
            traffic = None

    def finish():
        # after the result line is out: leave together, then take the communicator down (collective)
        if dist is not None:
            dist.barrier()
            if rccl_ok:
                try:
                    _lib.dist_finalize(L)
                except Exception as exc:  # noqa: BLE001
                    print(f"bench: dist_finalize: {exc}", file=sys.stderr, flush=True)

    if rank != 0:
        finish()
        return
    out = {
        "metric": "fine-grid LUP/s (RB-GS smoother updates per second of V-cycle time, fp64)",
        "value": 2 * ms * npts / (ms_per_step * 1e-3),
        "unit": "LUP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": workload, "global_points": int(npts), "parallelism": parallelism},
        "vcycles_per_s": 1.0 / (ms_per_step * 1e-3),
        "vcycle_without_metric_ms": vc_only_ms,
        "smoother": {"ms_per_sweep": sm_ms, "LUPs_per_s": npts / (sm_ms * 1e-3), "residual_ms": rs_ms,
                     "laplace_variant_ms_per_sweep": lap_ms,
                     "laplace_variant_LUPs_per_s": (npts / (lap_ms * 1e-3)) if lap_ms else None},
        "coarse_exact_sweeps_per_cycle": sweeps_per_cycle,
        "roofline": {"bound": "hbm", "achieved": achieved / world, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / world / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "level-1 RB-GS sweep (red+black, fused)" + ("" if world == 1 else ", per GPU, halo exchange included"),
                     "bytes_per_lup": BYTES_PER_LUP,
                     "sweeps_per_launch": 2, "algorithmic_bytes_per_launch": 2 * BYTES_PER_LUP * npts / world,
                     "avg_launch_ms": 2 * sm_ms,
                     "note": "one launch = two full red+black sweeps (temporal blocking); traffic = HBM bytes per "
                             "launch from rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, profiles/traffic_latest.json"},
        "rocm_stack": _lib.bound_libs(L),
    }
    if slab_check is not None:
        out["slab_check"] = slab_check
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out), flush=True)
    finish()


if __name__ == "__main__":
    main()
